NAME          RAW2COST
ROWS
 N  COST
 G  SUP1COST
 G  SUP2COST
 L  PURITY
 E  AMOUNT
COLUMNS
    SUP1      COST              3.60   SUP1COST          3.60
    SUP1      PURITY             .20   AMOUNT            1.00
    SUP2      COST              1.20   SUP2COST          1.20
    SUP2      PURITY             .40   AMOUNT            1.00
RHS
    RHS       AMOUNT          100.00   PURITY           35.00
ENDATA